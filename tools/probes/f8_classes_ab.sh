for cfg in "fp16x3h:" "fp16x3q:" "fp16x3q:1" "fp16f8:" "fp16f8:15" "fp16x3h:"; do
  p=${cfg%%:*}; m=${cfg##*:}
  if [ -n "$m" ]; then export DYT_F8_CLASSES=$m; else unset DYT_F8_CLASSES; fi
  python bench.py --precision $p --steps 8 --warmup 2 --no-cpu-baseline --no-parity-mode 2>&1 >/dev/null | grep timed | sed "s/^/[$cfg] /"
done
