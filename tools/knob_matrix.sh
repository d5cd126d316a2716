#!/bin/bash
# The GPU suite under every tuning-knob setting DESIGN.md section 4 lists, one line per setting with its pass count:
#   gpurun --timeout 1190 -- 'bash tools/knob_matrix.sh r03 [part]'   ->  gpurun_out/prof/r03_knobs[_part].txt
# The heavy full-size oracle tests (PGD loops at 81^2 x 250, 1025^2, Mimura T = 30: ~2 min of CPU oracle per pass, knob
# independent on the oracle side) run once, with the default setting; the other settings run the rest of the suite.
TAG=${1:-r03}
PART=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
F=$OUT/${TAG}_knobs$([ "$PART" = all ] || echo _$PART).txt
cd $REPO
SKIP='not pgd_loop and not self_selected and not mimura and not c2_solidbody_81x81 and not c3_schnakenberg and not c4_chemotaxis'
A=("FEMFCT_TILES=0" "FEMFCT_STRIPS=0" "FEMFCT_IMPLICIT=0" "FEMFCT_TILE4=0" "FEMFCT_TILE4=2" "FEMFCT_T4_DPP=0" "FEMFCT_GEOM_MASS=0" "FEMFCT_T4_XCD=1" "FEMFCT_FUSE_BUILD=0" "FEMFCT_FUSE_DUDT=0")
B=("FEMFCT_FUSE_FLUX=0" "FEMFCT_FUSE_END=0" "FEMFCT_DEEP_HALO=0" "FEMFCT_DEFER_CHECK=0" "FEMFCT_INLINE_OPS=0" "FEMFCT_LMASK=0" "FEMFCT_HALF_D=0" "FEMFCT_T4_WALK=0" "FEMFCT_T4_PAIR=0")
C=("FEMFCT_T4_INT=0" "FEMFCT_T4_SNAKE=0" "FEMFCT_T4_STAGGER_US=0" "FEMFCT_SPECIES_SOLVER=1" "FEMFCT_MESH_SOLVE=0" "FEMFCT_PREASSEMBLE=0" "FEMFCT_EXACT=1" "FEMFCT_STEPS_PER_GRAPH=1" "FEMFCT_PAIR_SHAPE=3")
case $PART in a) SET=("${A[@]}");; b) SET=("${B[@]}");; c) SET=("${C[@]}");; *) SET=("${A[@]}" "${B[@]}" "${C[@]}");; esac
: > $F
if [ "$PART" = a ] || [ "$PART" = all ]; then
  o=$(timeout -k 10 900 python -m pytest tests -m gpu -q -rf 2>&1)
  r=$(echo "$o" | tail -1)
  echo "(default setting, whole suite)          : $r" >> $F
  echo "$o" | grep "^FAILED" | sed 's/^/    /' >> $F
  echo "[knobs] default: $r"
fi
for kv in "${SET[@]}"; do
  o=$(env $kv timeout -k 10 600 python -m pytest tests -m gpu -q -rf -k "$SKIP" 2>&1)
  r=$(echo "$o" | tail -1)
  printf "%-40s: %s\n" "$kv" "$r" >> $F
  echo "$o" | grep "^FAILED" | sed 's/^/    /' >> $F
  echo "[knobs] $kv: $r"
done
echo "source_sha16 $(python -c 'import bench; print(bench.source_sha16())'); heavy oracle tests deselected for the non-default settings: -k \"$SKIP\"" >> $F
