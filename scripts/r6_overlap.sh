#!/bin/bash
# round 6: overlap pairs, static partition vs dynamic unit claims -- G4D_DYNAMIC_UNITS exists only with scripts/experiments/r6_dynamic_units.patch applied (not merged)
export PAIRS="fps_gather_grid:sa_xyz_mlp3_pair,fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#0,fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#1,fps_gather_grid:mlp_chain_interp_init,fps_gather_grid:mlp_chain_table_cells,fps_gather_grid:linear_interp_add,three_nn_pruned:mlp_chain_interp_init,three_nn_pruned:mlp_chain_table_cells,three_nn_pruned:group[mlp_chain_group_table+mlp_chain_group_table]#1"
for d in 0 1; do
  echo "### G4D_DYNAMIC_UNITS=$d"
  G4D_DYNAMIC_UNITS=$d python scripts/exp_overlap.py 240 fp32 2>&1 | grep -v amdgpu.ids
done
