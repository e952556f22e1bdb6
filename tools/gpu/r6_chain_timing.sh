# phase profile: chain vs single-block kernels (timing build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_chain_timing; rm -rf $O; mkdir -p $O
for B in 1024; do
MKWS_LIB=$GRAFT_REPO_ROOT/multilingual_kws_amd/lib/libmkws_hip_timing.so timeout 300 python tools/chain_timing.py $B > $O/out$B.log 2> $O/err$B.log; echo "rc=$?"
python - $O/err$B.log <<'PY'
import sys,re
lines=open(sys.argv[1]).read().splitlines()
# keep last pass of each setting
out=[];cur=[];setting=None
blocks=[]
for l in lines:
    if l.startswith('===='):
        if cur: blocks.append((setting,cur))
        setting=l;cur=[]
    elif l.startswith('---- pass'): cur=[]
    else: cur.append(l)
if cur: blocks.append((setting,cur))
for s,c in blocks[-2:]:
    print(s)
    for l in c:
        if 'chain-timing' in l or ('block-timing' in l and re.search(r'block(4b|4c|5a|5b|5c|6a) ',l)): print('  ',l[:230])
PY
done
