for c in 0 8 -4 -6 -8 -10 -12; do
  echo "== MANTA_VERIFY_ABC_C=$c"
  MANTA_VERIFY_ABC_C=$c timeout 120 python tools/verify_profile.py 100 2>&1 | tail -1
done
export MANTA_VERIFY_ABC_C=-8
tools/verify_timeline.sh 2>&1 | grep -v radix
