#!/bin/bash
# build a library variant for same-box A/B runs:  tools/build_variant.sh <name> "<extra compiler flags>" [objects to rebuild ...]
# -> groth16_amd/libg16_<name>.so (load with G16_LIB=...).  Objects not listed are reused from the main build.
set -e
cd "$(dirname "$0")/../groth16_amd/csrc"
name=$1; flags=$2; shift 2
objs=${@:-msm.o msm_bn254.o}
d=build_$name
mkdir -p $d
for o in api.o hosttest.o ntt.o witness_map.o msm.o synth.o setup.o serialize.o msm_bn254.o; do
  case " $objs " in *" $o "*) rm -f $d/$o ;; *) cp -p $o $d/$o; touch $d/$o ;; esac
done
make -j8 OBJDIR=$d OUT=../libg16_$name.so EXTRA="$flags" > $d/build.log 2>&1 || { tail -20 $d/build.log; exit 1; }
echo "built libg16_$name.so ($flags)"
