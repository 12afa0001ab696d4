// Typings of the JS host (index.js).  The job description is the reference's
// RenderJobSchema (client/src/renderer/RenderJobSchema.tsx:17-86) plus `sdfScene`.
export type Vec3 = [number, number, number];
export type UniformData = { type: "f" | "i" | "ui"; count: 1 | 2 | 3 | 4; data: number[] };
export type RenderJobLight =
  | { type: "point"; position: Vec3; color: Vec3; size: number }
  | { type: "sun"; direction: Vec3; color: Vec3 };
export interface Material {
  diffuse: Vec3; diffuse_cutoff: number; specular: Vec3; specular_cutoff: number; roughness: number; subsurface: number;
  subsurface_color: Vec3; ior: number; sky_color: Vec3; sky_floor: number; sky_scale: number; sky_radius: number; sky_axis: 0 | 1 | 2;
}
/** material values a shape of a composed scene can have of its own (RmSurface); missing ones = the reference defaults */
export interface Surface { diffuse?: Vec3; specular?: Vec3; roughness?: number; subsurface?: number; subsurface_color?: Vec3; ior?: number; }
export class Scene { kind: number; params: number[]; material: Material; key(): string; }
export class CsgScene extends Scene {
  constructor(material?: Partial<Material>);
  union(): this; smoothUnion(k: number): this; subtract(): this; intersect(): this; smoothSubtract(k: number): this; smoothIntersect(k: number): this;
  /** ABI 8: a ring, a capped cylinder (axis y) and a half space (unit normal) about `center` / through `point` */
  torus(center: Vec3, majorRadius: number, minorRadius: number, surface?: Surface): this; cylinder(center: Vec3, radius: number, halfHeight: number, surface?: Surface): this;
  plane(point: Vec3, normal: Vec3, surface?: Surface): this;
  /** `surface`: the material functions then depend on the position -- at a point, the values of the nearest shape */
  sphere(center: Vec3, radius: number, surface?: Surface): this; box(center: Vec3, halfExtents: Vec3, surface?: Surface): this;
  /** domain operators: transform the point the FOLLOWING primitives are evaluated at (sphere-grid.glsl's repeat; one level of tree.glsl's fold) */
  repeat(period: Vec3): this; fold(scale: number, offset: Vec3, angles?: Vec3): this;
  /** a shape whose distance term is another scene kind's estimator at p - center (RM_PRIM_KIND): a Mandelbulb cut by a box is table data */
  shape(scene: Scene, center?: Vec3, surface?: Surface): this;
  glsl(): string;
}
export class Mandelbulb extends Scene { constructor(power?: number, iterations?: number, bailout?: number, material?: Partial<Material>); }
export function singleSphere(center?: Vec3, radius?: number, material?: Partial<Material>): CsgScene;
export type RenderJobSchema = {
  reflectionIterationCounts: number[];
  normalDelta: number;
  sdfShaderSource: string;
  sdfScene: Scene;
  customShaderParameters: { [key: string]: UniformData };
  fogDensity: number;
  time: number;
  timeDelta: number;
  dof: { amount: number; distance: number; showFocusedArea: boolean };
  camera: {
    position: Vec3; motion: Vec3; rotation: ArrayLike<number>;
    mode: { type: "perspective"; fov: number } | { type: "orthographic"; size: number } | { type: "panoramic"; angleX: number; angleY: number };
  };
  render: {
    samplesPerPixel: number; exposure: number; subdivisions: number; width: number; height: number; frameid: number;
    blendWithPreviousFrameFactor: number; sampleYieldInterval: number; blendMode: "additive" | "mix"; renderMode: "full" | "preview";
  };
  lights: RenderJobLight[];
};
export type RenderJobFramebufferInfo = {
  width: number; height: number; frameid: number; download(plane?: 0 | 1 | 2): Float32Array;
  /** display.frag on the GPU: RGBA8, row 0 = bottom */
  present(samples: number): Uint8Array;
  /** canvas.toDataURL("image/png") of the presented frame (index.tsx:470-476) */
  toDataURL(samples: number): string;
};
export type ShaderError = { type: "vertex" | "fragment" | "program"; infoLog: string };
export class RenderJobContext {
  constructor(device?: number, flags?: number);
  fboCreate(width: number, height: number, frameid: number): RenderJobFramebufferInfo;
  fboDelete(width: number, height: number, frameid: number): void;
  close(): void;
}
/** A frame sharded over several GPUs by this one process: a native context per device, each holding one part of the frame's
 *  8-row stripes; `present(samples)` of its framebuffer set assembles the canvas on the first GPU (rm_present_sharded). */
export type ShardedFramebufferInfo = {
  width: number; height: number; frameid: number; sharded: true; dof: boolean; rows(): number[];
  present(samples: number, dof?: boolean): Uint8Array; toDataURL(samples: number): string;
  /** the present in two halves (rm_present_sharded_start / _finish): the frame travels while the next samples render; one at a time */
  startPresent(samples: number, dof?: boolean): void; finishPresent(): Uint8Array; pendingPresent: boolean;
};
export class ShardedRenderJobContext {
  constructor(devices?: number[], flags?: number, samplesInFlight?: number);
  fboCreate(width: number, height: number, frameid: number): ShardedFramebufferInfo;
  fboDelete(width: number, height: number, frameid: number): void;
  close(): void;
}
export type Present = (schema: RenderJobSchema, context: RenderJobContext, framebuffers: RenderJobFramebufferInfo, samplesSoFar: number) => void;
export function doRenderJob(schema: RenderJobSchema, context: RenderJobContext | ShardedRenderJobContext): Promise<
  (present: Present) => Generator<undefined, { success: boolean; why?: ShaderError | { type: "general"; infoLog: string } }, unknown>
>;
export function uniformsFromSchema(schema: RenderJobSchema, randNoise: [number, number]): ArrayBuffer;
export function halton(base: number): Generator<number, never, unknown>;
export function resetHalton(): void;
export function encodePng(rgba: Uint8Array, width: number, height: number, bottomUp?: boolean): Buffer;
export const RM: { [name: string]: number };
