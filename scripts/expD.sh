#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/expD; mkdir -p $O
run() { name=$1; k=$2; KNOCK="$k" python scripts/exp_knockout.py --steps 160 > $O/$name.json 2> $O/$name.err; python -c "import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('%-14s %8.1f us/step  %6.0f frames/s' % ('$name', d['ms_per_step']*1e3, d['value']))"; }
L=g4d_linear_f32; G=g4d_mlp_chain_group_table_f32
run base ""
run sa1mlp g4d_sa_xyz_mlp3_pair_f32
run sa2tab "$L#0"
run sa2mlp "$G#0,$G#1"
run sa2all "$L#0,$G#0,$G#1"
run sa3tab "$L#1"
run sa3mlp "$G#2,$G#3"
run sa3all "$L#1,$G#2,$G#3"
run fp3 "g4d_interp_concat_f32,$L#2,$L#3"
run fp2 "$L#4,g4d_mlp_chain_interp_init_f32"
run fp1 g4d_mlp_chain_table_f32
run lbs g4d_lbs_one_f32
run allmlp "g4d_sa_xyz_mlp3_pair_f32,$L,$G,g4d_interp_concat_f32,g4d_mlp_chain_interp_init_f32,g4d_mlp_chain_table_f32"
run allmlplbs "g4d_sa_xyz_mlp3_pair_f32,$L,$G,g4d_interp_concat_f32,g4d_mlp_chain_interp_init_f32,g4d_mlp_chain_table_f32,g4d_lbs_one_f32"
run base2 ""
