set -x
O=gpurun_out/r06d; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
export NRC_HPM_LIB=$PWD/nrc-hpm-renderer_amd/lib_stamps/libnrc_hpm.so
for a in "--config c2" "--config c2 --train 1" "--config c5" "--config c5 --train 1"; do timeout -k 10 200 python tools/xcd_balance.py $a 2>&1 | grep -v amdgpu.ids >> $O/xcd_balance.txt || exit 1; echo >> $O/xcd_balance.txt; done
echo done
