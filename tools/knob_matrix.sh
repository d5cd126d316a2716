#!/bin/bash
# The GPU suite under the tuning-knob settings DESIGN.md section 4 lists, one line per setting with its pass count:
#   gpurun --timeout 1190 -- 'bash tools/knob_matrix.sh r04 [part]'   ->  gpurun_out/prof/r04_knobs[_part].txt
# Part d: the default setting only (whole suite).  Part x: the settings in $KNOB_SET (whole suite each).  Parts f1..f5: the settings that switch bandwidth- or batch-regime kernels, the WHOLE suite each (heavy oracle tests
# included: those are the ones that meet the oracle at the config sizes).
# Parts a, b, c: the other settings, without the heavy full-size oracle tests (PGD loops at 81^2 x 250, 1025^2, Mimura
# T = 30: ~2 min of CPU oracle per pass); part a also runs the default setting with the whole suite.
TAG=${1:-r04}
PART=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
F=$OUT/${TAG}_knobs$([ "$PART" = all ] || echo _$PART).txt
cd $REPO
SKIP='not pgd_loop and not self_selected and not mimura and not c2_solidbody_81x81 and not c3_schnakenberg and not c4_chemotaxis'
A=("FEMFCT_TILES=0" "FEMFCT_STRIPS=0" "FEMFCT_IMPLICIT=0" "FEMFCT_TILE4=0" "FEMFCT_TILE4=2" "FEMFCT_T4_DPP=0" "FEMFCT_GEOM_MASS=0" "FEMFCT_T4_XCD=1" "FEMFCT_FUSE_BUILD=0" "FEMFCT_FUSE_DUDT=0")
B=("FEMFCT_FUSE_FLUX=0" "FEMFCT_FUSE_END=0" "FEMFCT_DEEP_HALO=0" "FEMFCT_DEFER_CHECK=0" "FEMFCT_INLINE_OPS=0" "FEMFCT_LMASK=0" "FEMFCT_HALF_D=0" "FEMFCT_T4_WALK=0" "FEMFCT_T4_PAIR=0")
C=("FEMFCT_GEOM_ROT=0" "FEMFCT_T4_INT=0" "FEMFCT_T4_SNAKE=0" "FEMFCT_T4_STAGGER_US=0" "FEMFCT_SPECIES_SOLVER=1" "FEMFCT_MESH_SOLVE=0" "FEMFCT_PREASSEMBLE=0" "FEMFCT_EXACT=1" "FEMFCT_STEPS_PER_GRAPH=1" "FEMFCT_MESH_STEP=0")
F1=("FEMFCT_MESH_STEP=0" "FEMFCT_T4_PAIR=0" "FEMFCT_T4_WALK=0")
F2=("FEMFCT_LMASK=0" "FEMFCT_HALF_D=0" "FEMFCT_T4_INT=0")
F3=("FEMFCT_INLINE_OPS=0" "FEMFCT_TILE4=0")
F4=("FEMFCT_TILE4=2" "FEMFCT_MESH_STEP_BATCH=8" "FEMFCT_GEOM_ROT=0")
F5=("FEMFCT_FORM_GROUPS=0" "FEMFCT_FUSE_END=0" "FEMFCT_MESH_SOLVE=0")
FULL=0
case $PART in a) SET=("${A[@]}");; b) SET=("${B[@]}");; c) SET=("${C[@]}");;
  f1) SET=("${F1[@]}"); FULL=1;; f2) SET=("${F2[@]}"); FULL=1;; f3) SET=("${F3[@]}"); FULL=1;; f4) SET=("${F4[@]}"); FULL=1;; f5) SET=("${F5[@]}"); FULL=1;; d) SET=(); FULL=1;; x) SET=(${KNOB_SET}); FULL=1;;
  *) SET=("${A[@]}" "${B[@]}" "${C[@]}");; esac
[ $FULL = 1 ] && SKIP=""
: > $F
if [ "$PART" = a ] || [ "$PART" = all ] || [ "$PART" = d ]; then
  o=$(timeout -k 10 900 python -m pytest tests -m gpu -q -rf 2>&1)
  r=$(echo "$o" | tail -1)
  echo "(default setting, whole suite)          : $r" >> $F
  echo "$o" | grep "^FAILED" | sed 's/^/    /' >> $F
  echo "[knobs] default: $r"
fi
for kv in "${SET[@]}"; do
  if [ -n "$SKIP" ]; then o=$(env $kv timeout -k 10 600 python -m pytest tests -m gpu -q -rf -k "$SKIP" 2>&1)
  else o=$(env $kv timeout -k 10 560 python -m pytest tests -m gpu -q -rf 2>&1); fi
  r=$(echo "$o" | tail -1)
  printf "%-40s: %s\n" "$kv" "$r" >> $F
  echo "$o" | grep "^FAILED" | sed 's/^/    /' >> $F
  echo "[knobs] $kv: $r"
done
if [ $FULL = 1 ]; then echo "build id $(python -c 'import bench; print(bench.source_sha16())'); whole -m gpu suite under each setting" >> $F
else echo "build id $(python -c 'import bench; print(bench.source_sha16())'); heavy oracle tests deselected for the non-default settings: -k \"$SKIP\"" >> $F; fi
