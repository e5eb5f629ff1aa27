def _unavailable(name):
    def ctor(*a, **k):
        raise NotImplementedError(f"torchvision.models.{name} is not available in the torchvision test shim "
                                  "(LPIPS needs the real package and downloaded weights)")
    return ctor


alexnet = _unavailable("alexnet")
squeezenet1_1 = _unavailable("squeezenet1_1")
vgg16 = _unavailable("vgg16")
