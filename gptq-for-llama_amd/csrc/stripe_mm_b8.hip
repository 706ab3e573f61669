// stripe16 small-batch MFMA kernel, 8-bit instantiations (stripe_mm.inc)
#define STRIPE_BITS 8
#include "stripe_mm.inc"
