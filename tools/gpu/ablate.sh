# phase ablations of the timing build (MKWS_ABLATE bit mask: the kernels skip those phases; results are wrong, the timing of the rest is honest)
cd $GRAFT_REPO_ROOT
for m in ${ABLATE_MASKS:-0 1 2 4 8 3}; do
  echo "== MKWS_ABLATE=$m"
  MKWS_ABLATE=$m WG_FILTER="${WG_FILTER:-mid-timing}" bash tools/gpu/wgtrace.sh | tail -n +2 | cut -c1-260
done
