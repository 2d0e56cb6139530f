# round 4: path-equivalence soaks with dormant rows carried on the device (measure/debug/loop_equiv_soak.py)
mkdir -p gpurun_out
( for env in "MAXD=30" "MAXD=30 AHEAD=1 PEEK=5" "REFINE=1 MAXD=30 AHEAD=1 PEEK=3" "SWITCH=1 MAXD=12 AHEAD=1 PEEK=2" "MAXD=6 SEED=7 AHEAD=1 PEEK=4"; do
    echo "== $env"; env $env timeout 150 python measure/debug/loop_equiv_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -4; echo "exit $?"
  done ) > gpurun_out/r04_carry_soak.log 2>&1
cut -c1-330 gpurun_out/r04_carry_soak.log
