// stripe16 small-batch MFMA kernel, 4-bit instantiations (stripe_mm.inc)
#define STRIPE_BITS 4
#include "stripe_mm.inc"
