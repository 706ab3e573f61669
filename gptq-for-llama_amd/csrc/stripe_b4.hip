// stripe_b4.hip -- the 4-bit instantiations of the stripe16 decode kernel (stripe_kernel.inc).
#define STRIPE_BITS 4
#include "stripe_kernel.inc"
