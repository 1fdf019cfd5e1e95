for arm in base xs0 new; do
  case $arm in
    base) export SMI_LIB=$PWD/sliders_conceptmod_amd/build/libsmi_r03.so; unset SMI_ATTN_XS;;
    xs0) unset SMI_LIB; export SMI_ATTN_XS=0;;
    new) unset SMI_LIB; unset SMI_ATTN_XS;;
  esac
  python -m pytest tests/test_engine_gpu.py -q -s -k "lora_gradients_match_oracle or rank8 or ragged" > gpurun_out/r4_gradbars_$arm.log 2>&1
  echo "== $arm"; grep -E "global|passed|failed|rel err" gpurun_out/r4_gradbars_$arm.log | grep -v "assert\|f\"" | cut -c1-160
done
