"""Drop-in for the reference's ``networks`` package on the distillation hot path:
pspnet_combine (student / teacher PSPNet), sagan_models (holistic discriminator), spectral
(spectral norm on the gfx950 kernel) and kd_model (NetModel, the step orchestrator)."""
