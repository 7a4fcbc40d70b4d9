#!/bin/bash
# k_decide_mr / k_apply_istft_mr launch-shape sweep (SG_MR_NT / SG_MR_TEAMS / SG_MR_FPW) for one n_fft: per-kernel event times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-1000}
for nt in 64 256; do for teams in 1 2 4 8; do for fpw in 1 4; do
  [ $nt = 256 ] && [ $teams != 1 ] && continue
  echo -n "n_fft=$N NT=$nt teams=$teams fpw=$fpw  "
  SG_MR_NT=$nt SG_MR_TEAMS=$teams SG_MR_FPW=$fpw python tools/prof_nfft.py $N 2>/dev/null | grep " stat " | python -c "
import sys,ast
for ln in sys.stdin:
    d=ast.literal_eval(ln[ln.index('{'):]); print({k[:22]:v for k,v in d.items() if 'apply' in k or 'stft_bits' in k or 'decide' in k or 'STFT' in k.upper()})"
done; done; done
