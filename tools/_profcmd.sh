# sourced by kstats.sh / pmc_kernel.sh: checks the command that goes behind `rocprofv3 ... --` and makes its script path absolute.
# On this pool the profiler's preloaded library initialises the GPU before the program starts (always with --pmc), and a process
# that has initialised the GPU must never exec another program: the PROGRAM ITSELF has to follow `--` — `python3 /abs/tool.py ...`
# or a binary — never env / bash -c / sh -c / taskset / numactl / a launcher, and never a `#!/usr/bin/env python3` script run directly.
# Knobs go into the environment of the calling script (export X=1 in front of it), as collect_profiles.sh does.
profcmd_check() {
  local first=${1%% *}
  case "$(basename -- "$first")" in
    env|bash|sh|dash|zsh|taskset|numactl|nice|timeout|stdbuf|time|torchrun)
      echo "$0: '$first' behind 'rocprofv3 --' is an exec after the profiler initialised the GPU: put the program itself there (python3 /abs/path/tool.py ...) and export knobs in this shell" >&2
      return 2;;
    *.py|*.sh)
      echo "$0: '$first' is a script with a shebang (an exec through env / the shell): write 'python3 $first ...'" >&2
      return 2;;
  esac
  return 0
}
# python3 tools/x.py a b  ->  python3 $ROOT/tools/x.py a b   (the profiler runs from /tmp)
profcmd_abs() {
  local root=$1; shift
  local out=() w
  for w in $*; do
    if [[ "$w" != /* && "$w" == *.py && -f "$root/$w" ]]; then out+=("$root/$w"); else out+=("$w"); fi
  done
  echo "${out[*]}"
}
