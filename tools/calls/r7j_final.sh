#!/bin/bash
# final validation of round 6 at the last kernel source (after the pair as two launches, the bounded waits, the K-sliced form): same script as r7d_final
exec bash tools/final_validation.sh r7j_final
