#!/bin/bash
# VALU instruction counts of the fit kernels under the timing-ablation flags (results are wrong when flags are set)
for f in "$@"; do
  export PWPP_DEBUG_FLAGS=$f
  echo "== PWPP_DEBUG_FLAGS=$f"
  bash $GRAFT_REPO_ROOT/tools/prof_pmc.sh abl$f SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE 2>&1 | grep -E "k_fit_w64|k_fit_srows"
done
