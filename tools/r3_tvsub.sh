#!/bin/bash
# chain sub-batch sweep: bench tv leg per SKM_TV_SUB value (one process each; same box, same call)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out
for sub in ${SUBS:-4096 1024 512 256 192 128 64}; do
  SKM_TV_SUB=$sub timeout -k 10 200 python bench.py --legs tv --no-cpu-baseline --steps 1 --warmup 0 > $O/r3_tvsub_$sub.json 2> $O/r3_tvsub_$sub.err || { echo "sub=$sub failed"; tail -3 $O/r3_tvsub_$sub.err; exit 1; }
  python - "$sub" "$O/r3_tvsub_$sub.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])['tv']
k=d['train_kernels_ms']; e=d['extract_kernels_ms']
ch=sum(k.get(n,0) for n in ('k_chol_diag','k_gemm_chol_panel','k_gemm_chol_update','k_gemm_trinv_a','k_gemm_trinv_b','k_gemm_xxt'))
print("   lbuild %.1f aacc %.1f aux %.1f cacc %.1f split %.1f split_e %.1f colsum %.1f | extract lbuild %.1f aux %.1f split %.1f"%(k.get('k_gemm_lbuild',0),k.get('k_gemm_aacc',0),k.get('k_gemm_aux',0),k.get('k_gemm_cacc',0),k.get('k_b3_split',0),k.get('k_b3_split_e',0),k.get('k_ehh_colsum',0),e.get('k_gemm_lbuild',0),e.get('k_gemm_aux',0),e.get('k_b3_split',0)))
print("sub=%5s train %.0f sess/s  iv %.0f /s  ivu %.0f /s | chain %.1f ms (diag %.1f panel %.1f update %.1f trinv %.1f+%.1f xxt %.1f) finish %.1f | extract: update %.1f panel %.1f diag %.1f solve %.1f"%(
 sys.argv[1], d['train_sessions_per_s'], d['ivectors_per_s'], d['ivectors_with_uncertainty_per_s'], ch, k['k_chol_diag'],k['k_gemm_chol_panel'],k['k_gemm_chol_update'],k['k_gemm_trinv_a'],k['k_gemm_trinv_b'],k['k_gemm_xxt'],k['k_tv_finish'],
 e['k_gemm_chol_update'],e['k_gemm_chol_panel'],e['k_chol_diag'],e.get('k_tv_solve',0)))
PY
done
