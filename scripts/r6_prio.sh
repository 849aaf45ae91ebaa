#!/bin/bash
export PAIRS="fps_gather_grid:sa_xyz_mlp3_pair,fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#0,fps_gather_grid:group[mlp_chain_group_table+mlp_chain_group_table]#1,fps_gather_grid:mlp_chain_interp_init,fps_gather_grid:mlp_chain_table_cells"
for o in 0 1 3; do
  echo "### G4D_FPS_PRIO=$o"
  G4D_FPS_PRIO=$o python scripts/exp_overlap.py 240 fp32 2>&1 | grep -v "amdgpu.ids\|Warn\|isinstance" | grep "^fps\|###\|alone fps"
done
