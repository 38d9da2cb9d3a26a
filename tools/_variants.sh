for v in "" u2 t512; do
  if [ -n "$v" ]; then export EHM_LIB=$PWD/explicit_hybrid_mpc_amd/lib/libehmpc_$v.so; else unset EHM_LIB; fi
  echo "=== variant [$v]"
  timeout 600 python tools/k2_check.py 2>&1 | grep -E "FAIL|ALL OK|SOME|decide_full=0|decide_full=1 bench partition gen 2" | tail -5
done
