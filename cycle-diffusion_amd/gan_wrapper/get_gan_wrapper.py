"""Plugin switch with the reference's kwarg remapping (model/gan_wrapper/get_gan_wrapper.py:3-30):
`[gan]` keys become constructor kwargs; `source_*` / `target_*` pick the source or target model."""


def get_gan_wrapper(args, target=False):
    kwargs = {}
    for kw, arg in args:
        if kw == "gan_type":
            continue
        if not kw.startswith("source_") and not kw.startswith("target_"):
            kwargs[kw] = arg
        elif target and kw.startswith("target_"):
            kwargs["source_" + kw[len("target_"):]] = arg
        elif not target and kw.startswith("source_"):
            kwargs[kw] = arg
    if args.gan_type == "DDPM_DDIM":
        from .ddpm_ddim_wrapper import DDPMDDIMWrapper
        return DDPMDDIMWrapper(**kwargs)
    if args.gan_type == "LatentDiffStochasticText":
        from .latent_text_wrapper import LatentDiffStochasticTextWrapper
        return LatentDiffStochasticTextWrapper(**kwargs)
    if args.gan_type == "SDStochasticText":
        from .latent_text_wrapper import SDStochasticTextWrapper
        return SDStochasticTextWrapper(**kwargs)
    if args.gan_type == "LatentDiffStochastic":
        from .latent_wrapper import LatentDiffStochasticWrapper
        return LatentDiffStochasticWrapper(**kwargs)
    raise ValueError(args.gan_type)
