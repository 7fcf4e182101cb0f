# tools/probe_shapes.sh (ON THE GPU BOX): the union factor of camera-ray packets by shape, 2^N passes x 64 >> N pixels (HR_TUNE plog=N: the selector's probe only)
for wl in c3 c2 c5 c3d terrain; do for n in 2 3 4 5 6; do
  echo -n "$wl plog=$n: "; HR_DEBUG_PIPE=1 HR_TUNE="plog=$n" timeout -k 10 200 python bench.py --quick --parity-seconds 0 --workload $wl --steps 40 2>&1 >/dev/null | grep "packet probe" | head -1 | sed 's/.*union/union/'
done; done
