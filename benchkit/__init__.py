"""Parts of bench.py that do not touch the CPU checker: launcher, synthetic inputs, timing fences, the workloads of the
timed region and the extra sections of the default line.  bench.py keeps the command line, the checker / CPU-baseline legs
and the assembly of the JSON line."""
