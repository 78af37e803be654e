#!/bin/bash
# development build for ONE padded state count into tools/exp/NAME.so (select it with TEHMM_HIP_LIB):
#   tools/devbuild.sh NAME [NT] [extra hipcc flags...]
name=$1; nt=${2:-36}; shift 2
mkdir -p /root/repo/tools/exp
/opt/rocm/bin/hipcc -DTEHMM_DEV_NT=$nt "$@" --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wall -Wno-unused-function -Wno-unused-lambda-capture \
  -o /root/repo/tools/exp/$name.so /root/repo/tehmm_amd/csrc/tehmm_hip.hip && echo built tools/exp/$name.so
