for cfg in "1 0" "8 8000" "4 16000"; do set -- $cfg
  export HP_STAGGER_PHASES=$1 HP_STAGGER_TICKS=$2
  echo "== phases=$1 ticks=$2"; python tools/scratch/trace2.py 2>&1 | grep -v amdgpu | grep "W= 2816"
done
