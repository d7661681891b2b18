# usage: bash tools/ab_rank.sh lib.so "ENV=..." ...   -> wall ms/frame of rank contexts (N = 8 rank 3, N = 2 rank 1) and of the whole frame under each environment
lib=$1; shift
for e in "$@"; do
  for wr in "8 3" "2 1" "1 0"; do
    env $e ZELDA_RENDER_LIB=$PWD/zeldaengine_amd/$lib timeout -k 10 100 python tools/rank_passes.py $wr 3 2>/dev/null | grep -E "wall|passes" | tr '\n' ' ' | cut -c1-330
    echo "   [$e]"
  done
done
