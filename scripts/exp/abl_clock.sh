#!/bin/bash
# effective clock of the GEMM launches in the ablation variants (GPU box): GRBM_GUI_ACTIVE / 8 XCDs / duration
export TMPDIR=/tmp GODE_AB_LIB=gan-ode_amd/lib/libgode_abl.so
for v in 0 -1 -3; do
  rm -rf /tmp/ablc$v
  GODE_IGEMM_STAGGER=$v rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/ablc$v -o t -- python3 scripts/exp/ab_igemm.py > /dev/null 2>&1
  echo "== stagger $v"
  python3 scripts/pmc_clock.py $(find /tmp/ablc$v -name '*counter_collection.csv' | head -1) | awk '{k=$NF" "$2; n[k]++; last[k]=$0} END{for(k in last) print last[k]}' | sort | head -20
done
