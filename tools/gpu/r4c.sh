# round 4: per-tile time breakdown (stamped debug build) of the ping-pong kernels by epilogue kind
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
make -C unimp_amd/csrc -j8 EXTRA=-DG3_STAMP OBJD=$PWD/build/obj_stamp OUT=$PWD/build/libunimp_hip_stamp.so > $O/make.log 2>&1; echo "make rc=$?" > $O/rc.txt
for v in pp256 pp256x; do
  for spec in "32768 2560 2560 $v plain 1" "32768 2560 2560 $v res 0" "32768 10240 2560 $v gelu2 0" "32768 10240 2560 $v aux 1" "32768 2560 10240 $v plain 1" "131584 1024 1024 $v res 0" "131584 4096 1024 $v bias 0"; do
    timeout 300 python tools/stamp_gemm3.py $spec >> $O/stamps.log 2>&1
  done
done
cat $O/stamps.log | grep -v amdgpu.ids
