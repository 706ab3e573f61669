#!/bin/bash
# final validation of round 6 with the sampling leg on the bench line (kernel sources unchanged since r7p_final): same script
exec bash tools/final_validation.sh r7z_final
