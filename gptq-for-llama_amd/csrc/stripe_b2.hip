// stripe_b2.hip -- the 2-bit instantiations of the stripe16 decode kernel (stripe_kernel.inc).
#define STRIPE_BITS 2
#include "stripe_kernel.inc"
