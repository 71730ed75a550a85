cd /tmp && export TMPDIR=/tmp
for shape in "512 512 64 3 1 1" "128 128 256 3 1 1"; do
tag=$(echo $shape | tr ' ' '_')
for kind in fwd wgrad; do
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /root/repo/gpurun_out/pmc3 -o f_${tag}_${kind} -- python /root/repo/tools/bench_one.py $shape 16 10 $kind > /root/repo/gpurun_out/pmc3_f_${tag}_${kind}.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /root/repo/gpurun_out/pmc3 -o w_${tag}_${kind} -- python /root/repo/tools/bench_one.py $shape 16 10 $kind > /root/repo/gpurun_out/pmc3_w_${tag}_${kind}.log 2>&1
done; done
ls /root/repo/gpurun_out/pmc3
