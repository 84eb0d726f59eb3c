#!/bin/bash
# Builds libctta_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libctta_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -fno-slp-vectorize: clang's SLP vectoriser pairs scalar fp32 work into v_pk_*_f32 with op_sel source swizzles, and
# `v_pk_fma_f32 ... op_sel:[0,1,0]` returns wrong low-lane results on these MI355X boxes while waves of ANOTHER kernel issue
# MFMAs on the same CU (tools/pk_hazard.py; LABNOTES.md 5) -- which is what two engine handles on two streams do.
# tools/check_isa.py (also a test) verifies that the built library contains no such form.
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-variable -Wno-pass-failed ${CTTA_EXTRA_FLAGS:-}"
mkdir -p build
pids=()
for f in conv_gemm_i1 conv_gemm_i2 conv_gemm_i3 conv_gemm_i4 conv_gemm_i5 conv_gemm_i6 conv_gemm_i7 conv_gemm_i8 conv_gemm_i9 api conv_gemm resunit ffn_fused clap_ops eval_ops wgrad_gemm norm_elem attention backward engine_unet engine_vae engine_t5 mel_frontend stft_loss; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ engine_common.h -nt build/$f.o ] || [ conv_epilogue.h -nt build/$f.o ] || { [[ $f == conv_gemm* ]] && [ conv_gemm_kernel.h -nt build/$f.o ]; } || [ engine_unet_train.h -nt build/$f.o ] || [ ../../include/ctta.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
# elementwise: no fp contraction so the Heun / EMA arithmetic rounds like the reference's unfused ops
if [ ! -f build/elementwise.o ] || [ elementwise.hip -nt build/elementwise.o ] || [ common.h -nt build/elementwise.o ] || [ ../../include/ctta.h -nt build/elementwise.o ]; then
  $HIPCC $FLAGS -ffp-contract=off -c elementwise.hip -o build/elementwise.o &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
