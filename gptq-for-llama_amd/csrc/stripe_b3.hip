// stripe16 decode kernel, 3-bit instantiations (stripe_kernel.inc)
#define STRIPE_BITS 3
#include "stripe_kernel.inc"
