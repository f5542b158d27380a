"""MI355X runtime: C-ABI binding, flat parameter storage, planned graph, executor, fused
optimizer, RCCL data parallelism."""
