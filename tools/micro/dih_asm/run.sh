#!/bin/bash
# bash tools/micro/dih_asm/run.sh <seconds> <module names...>: kernel <32> (and <0>) of each module beside the partner
cd "$(dirname "$0")"
S=$1; shift
K32=_Z10dih_kernelILi32EEvPK15HIP_vector_typeIfLj4EEPfi; K0=_Z10dih_kernelILi0EEvPK15HIP_vector_typeIfLj4EEPfi; K16=_Z10dih_kernelILi16EEvPK15HIP_vector_typeIfLj4EEPfi
for m in "$@"; do echo "== $m"; ./dih_driver $m.hsaco $S $K32 $K0 ${EXTRA_K}; done
