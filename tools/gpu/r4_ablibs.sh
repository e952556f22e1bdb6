cd $GRAFT_REPO_ROOT
export KFILTER=chain
L=multilingual_kws_amd/lib
bash tools/gpu/ablibs.sh - $L/libmkws_hip_wq1.so - $L/libmkws_hip_wq1.so - $L/libmkws_hip_wq1.so
