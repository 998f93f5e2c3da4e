for e in "AG_NOHALF=1" "AG_X=1"; do
  echo "== $e"
  for cfg in "conv 256 512 7 2 3 256 0" "conv 256 512 7 2 3 256 1" "conv 128 256 7 2 3 512 0" "conv 128 256 7 2 3 512 1" "conv 64 128 7 2 3 1024 1" "conv 64 128 7 2 3 1024 0"; do
    env $e timeout -k 10 60 python tools/prof_conv.py $cfg | tail -1 || exit 1
  done
done
