for m in 128 1024; do
  GEOSSL_PAINN_NO_SPLIT=1 GEOSSL_BUCKET_NO_HEADROOM=1 python tools/experiments/painn_forms.py C $m 2>&1 | tail -1
  GEOSSL_PAINN_NO_SPLIT=1 python tools/experiments/painn_forms.py C $m 2>&1 | tail -1
  python tools/experiments/painn_forms.py C $m 2>&1 | tail -1
  GEOSSL_PAINN_MMA_CAP=44 python tools/experiments/painn_forms.py C $m 2>&1 | tail -1
done
