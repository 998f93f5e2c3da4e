# usage: bash tools/_ab_layers.sh lib1.so lib2.so ...  (per-layer bf16 timings, same box)
for lib in "$@"; do
  cp $lib audiogan_amd/libaudiogan_hip.so
  echo "== $lib"
  timeout -k 10 200 python tools/prof_layers.py 64 bf16 2>&1 | grep -v amdgpu.ids | awk '$2=="fwd"||$2=="bwd-x"{printf "%s/%s %.1f | ", $1,$2,$3} END{print ""}' || exit 1
done
