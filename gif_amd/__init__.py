"""gif_amd — MI355X-native hot path of ParthaEth/GIF (StyleGAN2 G/D step + FLAME mesh rasteriser).

Drop-in classes (same constructor / forward signatures and state_dict keys as the reference):
    gif_amd.generator.StyledGenerator        <- model/stg2_generator.py
    gif_amd.discriminator.Discriminator      <- model/stg2_discriminator.py
    gif_amd.standard_rasterize               <- my_utils/standard_rasterize_cuda (pybind module + visibility.py)
    gif_amd.losses                           <- loss_functions/losses.py (R1, path length)
    gif_amd.train_step                       <- train.py loop body, DataParallel -> one process per GPU + RCCL
The compute is in gif_amd/libgif_hip.so (hand-written HIP for gfx950, C ABI in include/gif_hip.h).
"""
__all__ = ["generator", "discriminator", "layers", "functional", "ops", "standard_rasterize", "losses", "train_step"]
