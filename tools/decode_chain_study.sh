# Why a decode-step skinny GEMM costs ~6 us for < 1 us of HBM time: one gpurun call, everything into gpurun_out/decode_chain_study.txt
#   1. tools/ubench/kernarg_chain{,_preload}: struct-by-value vs scalar (preloaded) kernel arguments in a chain of dependent launches
#   2. tools/ubench/dispatch_rate: how fast the blocks of one short kernel start (grid 192..1024, 256 / 512 threads, with registers, barrier, 2-D grid)
#   3. tools/skinny_chain_probe.py: node-to-node time of dependent QKV / O / wi / wo launches (hot weights, HBM weights, four shapes)
#   4. tools/skinny_stamps.py on the SKINNY_STAMPS build: the phases inside the kernel and the entry time of every block
#   5. tools/decode_ab.py: block-shape sweeps on the real greedy decode loop
cd $GRAFT_REPO_ROOT
[ -f tools/libvid2seq_hip_stamps.so ] || { echo "build tools/libvid2seq_hip_stamps.so first (tools/build_skinny_stamps.sh, in the build container)"; exit 1; }
O=gpurun_out/decode_chain_study.txt
{
echo "== 1. kernel arguments (us per dependent node, replayed graph of 1000 nodes)"
for b in kernarg_chain kernarg_chain_preload; do echo "-- $b"; tools/ubench/$b; done
echo; echo "== 2. workgroup dispatch"
tools/ubench/dispatch_rate
echo; echo "== 3. dependent skinny GEMMs, node to node"
python tools/skinny_chain_probe.py 2>&1 | grep -v amdgpu.ids
echo; echo "== 4. phases inside the kernel (SKINNY_STAMPS build)"
V2S_LIB=tools/libvid2seq_hip_stamps.so python tools/skinny_stamps.py 2>&1 | grep -v amdgpu.ids
echo; echo "== 5. block shapes on the greedy decode loop (B = 64, 64 steps; 1<waves><mt><nt>; 20000 + n: at most n blocks with one row fragment per block)"
ROUNDS=5 python tools/decode_ab.py gemm_skinny=1 gemm_skinny=1821 gemm_skinny=1812 gemm_skinny=1822 gemm_skinny=1411 gemm_skinny=20600 gemm_skinny=20400 gemm_skinny=20300 gemm_skinny=20200 2>&1 | grep -v amdgpu.ids | tail -9
} > $O 2>&1
tail -5 $O
