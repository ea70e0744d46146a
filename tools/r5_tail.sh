# round 5: the last four convolutions of the extras chain (behind the 3x3 map) in one launch (WZ_TAIL_FUSE=2, development knob) against four launches
mkdir -p gpurun_out/r05j
for b in 8 1; do for v in "WZ_TAIL_FUSE=0" "WZ_TAIL_FUSE=2" "WZ_TAIL_FUSE=0" "WZ_TAIL_FUSE=2" "WZ_TAIL_FUSE=1"; do
  echo "== batch $b $v"
  env $v timeout 150 python tools/stage_table.py --robust --throughput --batch $b --only layer_19 2>&1 | grep -E "layer_19|throughput|^sum"
done; done > gpurun_out/r05j/tail.txt 2>&1
cat gpurun_out/r05j/tail.txt
