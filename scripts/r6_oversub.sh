#!/bin/bash
# round 6 experiment: static partition with MORE workgroups than fit an empty chip (the dispatcher balances) vs the default, next to the sampling launch
export PAIRS="fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#0,fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#1,three_nn_pruned:group[mlp_chain_group_table+mlp_chain_group_table]#1"
for o in 1 2 4 8; do
  echo "### G4D_SA_TABLE_OVERSUB=$o (G4D_DYNAMIC_UNITS=0)"
  G4D_DYNAMIC_UNITS=0 G4D_SA_TABLE_OVERSUB=$o python scripts/exp_overlap.py 240 fp32 2>&1 | grep -v "amdgpu.ids\|Warn\|isinstance" | grep "group\|###"
done
