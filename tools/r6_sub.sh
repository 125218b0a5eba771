for a in "--steps 20 --warmup 3" "--op AND_NOT --terms 4 --required 2 --steps 6 --warmup 1" "--op AND_MAYBE --terms 4 --required 2 --steps 6 --warmup 1" "--op PHRASE --topk 10 --steps 10 --warmup 2" "--op PHRASE --topk 10 --replay frozen --steps 10 --warmup 2" "--replay count --steps 10 --warmup 2"; do
  echo "== $a"
  bash tools/ab_run.sh "$a --no-other-configs --no-hook-parity --ref-docs 0" newk oldk newk oldk
done
