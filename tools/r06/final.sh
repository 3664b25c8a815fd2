#!/bin/bash
# the round's evidence stamp: GPU suite, bench lines (C3 default line, C2, C5, stage 1, 8 clips, latency, LM), rocprofv3 kernel stats, PMC passes
bash tools/final_round.sh r06 2>&1 | tail -40
bash tools/pmc_collect.sh r06 2>&1 | tail -5
