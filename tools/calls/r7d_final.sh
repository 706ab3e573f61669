#!/bin/bash
# final validation of round 6 at this source: smoke, the whole GPU suite, PMC traffic pass, the bench line, rocprofv3 kernel stats (pass, decode engine at 1 / 4 / 16 rows, prefill), tile GEMM counters
exec bash tools/final_validation.sh r7d_final
