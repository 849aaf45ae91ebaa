#!/bin/bash
# What each launch (group) of the cfg2 step costs the bench's mix (round 4: 30 steps per call, 3 calls in flight): bench.py with those C-ABI calls skipped (scripts/exp_knockout.py,
# diagnostic -- not a throughput number); the drop in us/step against `base` is the marginal cost.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; k=$2; KNOCK="$k" python scripts/exp_knockout.py --steps 120 --min-seconds 1.5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %8.1f us/step  %6.0f frames/s' % ('$name', d['ms_per_step']*1e3, d['value']))"; }
L=g4d_linear_f32; G=g4d_mlp_chain_group_table_f32
run "base (nothing skipped)" ""
run "without SA1 MLP pair" g4d_sa_xyz_mlp3_pair_f32
run "without SA2 table" "$L#0"
run "without SA2 MLPs (pair launch)" "$G#0,$G#1"
run "without SA3 table" "$L#1"
run "without SA3 MLPs (pair launch)" "$G#2,$G#3"
run "without FP3 (3 launches)" "g4d_interp_concat_f32,$L#2,$L#3"
run "without FP2 (table + chain)" "$L#4,g4d_mlp_chain_interp_init_f32"
run "without FP1 + head" "g4d_mlp_chain_table_cells_f32"
run "without lbs()" g4d_lbs_one_f32
run "without every shared-MLP launch" "g4d_sa_xyz_mlp3_pair_f32,$L,$G,g4d_interp_concat_f32,g4d_mlp_chain_interp_init_f32,g4d_mlp_chain_table_cells_f32"
run "... and without lbs()" "g4d_sa_xyz_mlp3_pair_f32,$L,$G,g4d_interp_concat_f32,g4d_mlp_chain_interp_init_f32,g4d_mlp_chain_table_cells_f32,g4d_lbs_one_f32"
run "base again" ""
