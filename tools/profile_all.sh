#!/bin/bash
# The round's six profile sets in one gpurun call: fp32 headline, bf16, Path B, Path B on the x250 up-sampled input, f32_split, Path B f32_split.
#   gpurun --timeout 2400 -- tools/profile_all.sh ;  then here: tools/profile_all.sh summarize r04
cd "$(dirname "$0")/.."
if [ "$1" = summarize ]; then
  T=${2:-r04}
  python3 tools/summarize_profiles.py ${T} prof fp32
  python3 tools/summarize_profiles.py ${T}_bf16 prof_bf16 bf16
  python3 tools/summarize_profiles.py ${T}_pathB prof_pathB pathB
  python3 tools/summarize_profiles.py ${T}_pathB_pad250 prof_pathB_pad250 pathB --workload spectrogram --num-pad-frames 250
  python3 tools/summarize_valu.py ${T} ${T}
  python3 tools/summarize_profiles.py ${T}_f32split prof_f32split f32_split
  python3 tools/summarize_profiles.py ${T}_pathB_f32split prof_pathB_f32split pathB_f32_split
else
  tools/profile_round.sh prof
  tools/profile_round.sh prof_bf16 --mfma bf16
  tools/profile_round.sh prof_pathB --workload spectrogram
  SAR_VALU_PASS=1 tools/profile_round.sh prof_pathB_pad250 --workload spectrogram --num-pad-frames 250
  tools/profile_round.sh prof_f32split --mfma f32_split
  tools/profile_round.sh prof_pathB_f32split --workload spectrogram --mfma f32_split
fi
