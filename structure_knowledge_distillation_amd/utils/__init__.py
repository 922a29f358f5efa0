"""Drop-in for the reference's ``utils`` package on the distillation hot path: criterion (the six
loss classes), utils (similarity helpers + tuple-string builders) and parallel (one-process-per-GPU
data parallelism over RCCL behind the reference's wrapper names)."""
