// stripe16 small-batch MFMA kernel, 3-bit instantiations (stripe_mm.inc: K slices, prescale where the group is smaller than a row block)
#define STRIPE_BITS 3
#include "stripe_mm.inc"
