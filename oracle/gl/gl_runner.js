// Software-GL runner for the reference's GLSL, loaded by Kaleido's bundled
// HeadlessChrome (SwiftShader WebGL2) as a stand-in for "plotly.js".
//
// TEST INFRASTRUCTURE ONLY. This file contains no reference source: the
// shader text is read from /root/reference at run time by glref.py and handed
// over inside the job description. It only reproduces the GL calls the
// reference's executor issues around the draw
// (client/src/renderer/RenderJobExecutor.tsx:195-326 and
// client/src/renderer/LoadRenderJobContext.tsx:50-160): two framebuffers of
// three float attachments, "previous" textures on units 0..2, uniforms, one
// full-screen draw per sample, then curr becomes prev.
(function () {
  function b64(bytes) {
    var s = "";
    var chunk = 0x8000;
    for (var i = 0; i < bytes.length; i += chunk) {
      s += String.fromCharCode.apply(null, bytes.subarray(i, i + chunk));
    }
    return btoa(s);
  }

  function compile(gl, type, src) {
    var sh = gl.createShader(type);
    gl.shaderSource(sh, src);
    gl.compileShader(sh);
    if (!gl.getShaderParameter(sh, gl.COMPILE_STATUS)) {
      throw new Error("compile: " + gl.getShaderInfoLog(sh));
    }
    return sh;
  }

  function makeTargets(gl, w, h) {
    // attachment formats as LoadRenderJobContext.tsx:57-119
    var fmts = [gl.RGBA32F, gl.RGBA16F, gl.RGBA16F];
    var tex = [];
    var fbo = gl.createFramebuffer();
    gl.bindFramebuffer(gl.FRAMEBUFFER, fbo);
    for (var i = 0; i < 3; i++) {
      var t = gl.createTexture();
      gl.bindTexture(gl.TEXTURE_2D, t);
      gl.texStorage2D(gl.TEXTURE_2D, 1, fmts[i], w, h);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MIN_FILTER, gl.NEAREST);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_MAG_FILTER, gl.NEAREST);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_S, gl.REPEAT);
      gl.texParameteri(gl.TEXTURE_2D, gl.TEXTURE_WRAP_T, gl.REPEAT);
      gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0 + i, gl.TEXTURE_2D, t, 0);
      tex.push(t);
    }
    gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1, gl.COLOR_ATTACHMENT2]);
    gl.clearColor(0, 0, 0, 0);
    gl.clear(gl.COLOR_BUFFER_BIT);
    return { fbo: fbo, tex: tex };
  }

  function setUniform(gl, prog, name, u) {
    var loc = gl.getUniformLocation(prog, name);
    if (loc === null) return;
    if (u.kind === "mat4") { gl.uniformMatrix4fv(loc, false, new Float32Array(u.data)); return; }
    var fn = "uniform" + u.count + u.type + "v";
    gl[fn](loc, u.type === "f" ? new Float32Array(u.data) : new Int32Array(u.data));
  }

  function run(job) {
    var w = job.width, h = job.height;
    var canvas = document.createElement("canvas");
    canvas.width = w; canvas.height = h;
    var gl = canvas.getContext("webgl2", { preserveDrawingBuffer: true });
    if (!gl) throw new Error("no webgl2");
    if (!gl.getExtension("EXT_color_buffer_float")) throw new Error("no EXT_color_buffer_float");
    var info = {
      version: gl.getParameter(gl.VERSION),
      glsl: gl.getParameter(gl.SHADING_LANGUAGE_VERSION),
      renderer: gl.getParameter(gl.RENDERER),
      cores: navigator.hardwareConcurrency,
      ua: navigator.userAgent
    };
    var prog = gl.createProgram();
    gl.attachShader(prog, compile(gl, gl.VERTEX_SHADER, job.vert));
    gl.attachShader(prog, compile(gl, gl.FRAGMENT_SHADER, job.frag));
    gl.linkProgram(prog);
    if (!gl.getProgramParameter(prog, gl.LINK_STATUS)) throw new Error("link: " + gl.getProgramInfoLog(prog));
    gl.useProgram(prog);

    // full-screen quad, LoadRenderJobContext.tsx:257-266
    var vao = gl.createVertexArray();
    gl.bindVertexArray(vao);
    var vbo = gl.createBuffer();
    gl.bindBuffer(gl.ARRAY_BUFFER, vbo);
    gl.bufferData(gl.ARRAY_BUFFER, new Float32Array([-1, -1, 1, -1, -1, 1, 1, 1, -1, 1, 1, -1]), gl.STATIC_DRAW);
    var loc = gl.getAttribLocation(prog, "vertex_position");
    gl.vertexAttribPointer(loc, 2, gl.FLOAT, false, 8, 0);
    gl.enableVertexAttribArray(loc);

    var prev = makeTargets(gl, w, h);
    var curr = makeTargets(gl, w, h);
    gl.viewport(0, 0, w, h);
    if (job.init_prev0) {
      // per-pixel inputs for harness mains: preload the RGBA32F "previous colour" texture
      var raw = atob(job.init_prev0);
      var bytes = new Uint8Array(raw.length);
      for (var bi = 0; bi < raw.length; bi++) bytes[bi] = raw.charCodeAt(bi);
      gl.bindTexture(gl.TEXTURE_2D, prev.tex[0]);
      gl.texSubImage2D(gl.TEXTURE_2D, 0, 0, 0, w, h, gl.RGBA, gl.FLOAT, new Float32Array(bytes.buffer));
    }

    function draw(uniforms) {
      for (var i = 0; i < 3; i++) {
        gl.activeTexture(gl.TEXTURE0 + i);
        gl.bindTexture(gl.TEXTURE_2D, prev.tex[i]);
      }
      gl.bindFramebuffer(gl.DRAW_FRAMEBUFFER, curr.fbo);
      gl.drawBuffers([gl.COLOR_ATTACHMENT0, gl.COLOR_ATTACHMENT1, gl.COLOR_ATTACHMENT2]);
      setUniform(gl, prog, "previousColor", { type: "i", count: 1, data: [0] });
      setUniform(gl, prog, "previousNormalAndDofRadius", { type: "i", count: 1, data: [1] });
      setUniform(gl, prog, "previousAlbedoAndDepth", { type: "i", count: 1, data: [2] });
      for (var name in uniforms) setUniform(gl, prog, name, uniforms[name]);
      gl.drawArrays(gl.TRIANGLES, 0, 6);
      // the reference copies curr -> prev with a blit pass
      // (RenderJobExecutor.tsx:301-326); exchanging the two sets is the same
      // state for the next draw, because the raymarcher writes every texel.
      var t = prev; prev = curr; curr = t;
    }

    function finish() {
      gl.bindFramebuffer(gl.READ_FRAMEBUFFER, prev.fbo);
      gl.readBuffer(gl.COLOR_ATTACHMENT0);
      var px = new Float32Array(4);
      gl.readPixels(0, 0, 1, 1, gl.RGBA, gl.FLOAT, px);
    }

    var timings = [];
    var base = job.uniforms || {};
    var draws = job.draws || [{}];
    for (var d = 0; d < draws.length; d++) {
      var u = {};
      for (var k in base) u[k] = base[k];
      for (var k2 in draws[d]) u[k2] = draws[d][k2];
      var t0 = performance.now();
      draw(u);
      if (job.time) { finish(); timings.push(performance.now() - t0); }
    }

    var planes = job.read || [0];
    var out = {};
    gl.bindFramebuffer(gl.READ_FRAMEBUFFER, prev.fbo);
    for (var p = 0; p < planes.length; p++) {
      gl.readBuffer(gl.COLOR_ATTACHMENT0 + planes[p]);
      var buf = new Float32Array(w * h * 4);
      gl.readPixels(0, 0, w, h, gl.RGBA, gl.FLOAT, buf);
      out["plane" + planes[p]] = b64(new Uint8Array(buf.buffer));
    }
    var display = null;
    if (job.display) {
      // present pass (client/src/index.tsx:25-59): the display program samples the
      // three accumulated textures and writes the RGBA8 canvas image
      var dprog = gl.createProgram();
      gl.attachShader(dprog, compile(gl, gl.VERTEX_SHADER, job.display.vert));
      gl.attachShader(dprog, compile(gl, gl.FRAGMENT_SHADER, job.display.frag));
      gl.linkProgram(dprog);
      if (!gl.getProgramParameter(dprog, gl.LINK_STATUS)) throw new Error("link display: " + gl.getProgramInfoLog(dprog));
      gl.useProgram(dprog);
      var dloc = gl.getAttribLocation(dprog, "vertex_position");
      gl.vertexAttribPointer(dloc, 2, gl.FLOAT, false, 8, 0);
      gl.enableVertexAttribArray(dloc);
      var names = ["color", "normalAndDofRadiusTex", "albedoAndDepthTex"];
      for (var t = 0; t < 3; t++) {
        gl.activeTexture(gl.TEXTURE0 + t);
        gl.bindTexture(gl.TEXTURE_2D, prev.tex[t]);
        setUniform(gl, dprog, names[t], { type: "i", count: 1, data: [t] });
      }
      setUniform(gl, dprog, "brightness", { type: "f", count: 1, data: [job.display.brightness] });
      var dfbo = gl.createFramebuffer();
      var dtex = gl.createTexture();
      gl.activeTexture(gl.TEXTURE0 + 3);
      gl.bindTexture(gl.TEXTURE_2D, dtex);
      gl.texStorage2D(gl.TEXTURE_2D, 1, gl.RGBA8, w, h);
      gl.bindFramebuffer(gl.FRAMEBUFFER, dfbo);
      gl.framebufferTexture2D(gl.FRAMEBUFFER, gl.COLOR_ATTACHMENT0, gl.TEXTURE_2D, dtex, 0);
      gl.drawBuffers([gl.COLOR_ATTACHMENT0]);
      gl.viewport(0, 0, w, h);
      gl.drawArrays(gl.TRIANGLES, 0, 6);
      gl.readBuffer(gl.COLOR_ATTACHMENT0);
      var bytes8 = new Uint8Array(w * h * 4);
      gl.readPixels(0, 0, w, h, gl.RGBA, gl.UNSIGNED_BYTE, bytes8);
      display = b64(bytes8);
    }
    return { ok: true, info: info, timings_ms: timings, planes: out, display: display, err: gl.getError() };
  }

  window.Plotly = {
    version: "2.0.0",
    toImage: function (fig, opts) {
      return new Promise(function (resolve) {
        try {
          resolve(JSON.stringify(run(fig.layout.oracle)));
        } catch (e) {
          resolve(JSON.stringify({ ok: false, error: String(e && e.message ? e.message : e) }));
        }
      });
    }
  };
})();
