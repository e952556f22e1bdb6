# dense layers of a 256-clip (and 512-clip) handle under forced tile shapes (MKWS_GEMM_FORCE = "Mmax,MT,NT,SK")
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_gemm256; rm -rf $O; mkdir -p $O
for B in 256 512; do
for f in default 1,2,1 1,2,2 1,2,4 1,4,1 1,4,2 2,2,1 2,2,2 2,4,1 1,1,1 1,1,2; do
  if [ $f = default ]; then unset MKWS_GEMM_FORCE; else export MKWS_GEMM_FORCE=$B,$f; export MKWS_GEMM_FORCE_MIN=$B; fi
  timeout 120 python tools/kernel_table.py $B 20 dense > $O/t.txt 2>&1
  echo "B=$B force=$f: $(awk '/pass 1/{f=1} f && /dense/{s+=$(NF-1); printf "%s=%s ", $1, $(NF-1)} END{printf " sum=%.1f", s}' $O/t.txt)  $(grep 'pass 1' $O/t.txt)"
done; done
