// stripe_b8.hip -- the 8-bit instantiations of the stripe16 decode kernel (stripe_kernel.inc).
#define STRIPE_BITS 8
#include "stripe_kernel.inc"
