#!/bin/bash
# final validation of round 6 (kernel sources unchanged since r7p_final; host side: greedy + sampling generate paths): same script
exec bash tools/final_validation.sh r7x_final
