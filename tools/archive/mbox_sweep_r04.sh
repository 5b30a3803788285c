#!/bin/bash
# k_fit_mbox configurations (streaming waves per workgroup : waves per SIMD) on the bench workload
for cfg in "3:4" "7:4" "7:3" "5:3" "11:3" "15:4"; do
  echo "== PWPP_MBOX_CFG=$cfg"
  PWPP_MBOX_CFG=$cfg tools/plan_sweep_r04.sh "W16:1023,M64.4:65535"
done
tools/plan_sweep_r04.sh ""
