# usage: bash tools/cmp_conv.sh abl/old.so abl/new.so   (same box, back to back)
for lib in "$@"; do
  cp $lib audiogan_amd/libaudiogan_hip.so
  echo "== $lib"
  for cfg in "conv 256 512 7 2 3 256 0" "conv 256 512 7 2 3 256 1" "conv 128 256 7 2 3 512 0" "conv 128 256 7 2 3 512 1" "conv 64 128 7 2 3 1024 1" "convt 64 32 8 4 2 2048 0" "convt 64 32 8 4 2 2048 1" "conv 49 64 9 4 4 8192 0" "conv 49 64 9 4 4 8192 1" "convt 128 16 16 8 4 1024 0"; do
    timeout -k 10 60 python tools/prof_conv.py $cfg | tail -1 || exit 1
  done
done
