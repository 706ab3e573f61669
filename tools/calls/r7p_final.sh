#!/bin/bash
# final validation of round 6 at the final source (csrc frozen after this call): same script as r7d_final / r7j_final
exec bash tools/final_validation.sh r7p_final
