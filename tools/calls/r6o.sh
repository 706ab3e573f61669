#!/bin/bash
# persistent LDS-DMA engine with free hand-offs (tools/persistlab.hip): the ceiling of the structure against the product's 128 launches
# (sweeps: build with -DLAG= -DCH= -DPRIO= -DROT=)
O=gpurun_out/r6o; mkdir -p $O
( timeout 300 tools/persistlab 32 20; timeout 300 tools/persistlab 32 20 ) > $O/persistlab_default.txt 2>&1; cat $O/persistlab_default.txt | cut -c1-420
