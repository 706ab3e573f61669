#!/bin/bash
# final validation after the batched greedy-generate path (kernel sources unchanged since r7p_final): same script
exec bash tools/final_validation.sh r7t_final
