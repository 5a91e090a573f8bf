cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
VARIANTS="100 100 100 100" bash tools/sweep_libs.sh fir255_dec4_2p28 > gpurun_out/r02n/edge_ab.txt 2>&1
cat gpurun_out/r02n/edge_ab.txt | cut -c1-80
cp qo-100-tools_amd/libif_fir.so /tmp/orig.so
cp qo-100-tools_amd/libif_fir_ab_edge.so qo-100-tools_amd/libif_fir.so
PMC_TIMEOUT=120 bash tools/pmc_variants.sh fir255_dec4_2p28 FETCH_SIZE -- 100 > gpurun_out/r02n/edge_fetch.txt 2>&1
cp /tmp/orig.so qo-100-tools_amd/libif_fir.so
PMC_TIMEOUT=120 bash tools/pmc_variants.sh fir255_dec4_2p28 FETCH_SIZE -- 100 > gpurun_out/r02n/base_fetch.txt 2>&1
grep -A2 "fir_fft" gpurun_out/r02n/edge_fetch.txt gpurun_out/r02n/base_fetch.txt | grep FETCH
