// Fast build of the kernels: FMA contraction, v_rcp/v_sqrt/v_sin/v_cos/v_exp/v_log
// at hardware rate, trig-free power-8 Mandelbulb.  The random stream is the
// same bits as in the parity build (rm_device.hpp rm_tan / Rng).
#define RM_NS rm_fast
#define RM_FAST 1
#include "rm_device.hpp"
#include "rm_kernels.inc"
