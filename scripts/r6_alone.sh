#!/bin/bash
# the units of one 240-cloud call, each alone (us, median of 5)
PAIRS="none:none" python scripts/exp_overlap.py 240 ${1:-fp32} 2>&1 | grep "^alone\|^sum"
