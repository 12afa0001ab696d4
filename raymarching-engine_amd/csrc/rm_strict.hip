// Parity build of the kernels: compiled with -ffp-contract=off, IEEE divide and sqrt.
#define RM_NS rm_strict
#include "rm_device.hpp"
#include "rm_kernels.inc"
