cd /root/repo
mkdir -p gpurun_out
{
echo "== all"; python tools/wb_bench.py --reps 30 2>&1 | tail -4
for i in 0 1 2 3 4 5 6 7 8 9 10; do echo "== layer $i"; python tools/wb_bench.py --reps 30 --layers $i 2>&1 | tail -4; done
} > gpurun_out/wb_layers.txt 2>&1
