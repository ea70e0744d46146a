"""Device enumeration for the MI355X family, beside `edge_tpus()` / `cuda_gpus()` / `cpus()` of
`watsor/detection/devices.py:4-96`."""
import os


def hip_gpus():
    """Yields all available AMD GPU devices, if not subject for the following conditions
    (same conventions as `cuda_gpus()`, watsor/detection/devices.py:28-77):

    - set HIP_DEVICE environmental variable to specific device ID to use only the given device.

    - the default device can be specified in the file ~/.hip_device

    - additionally, HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restrict what the runtime exposes.
    """
    try:
        from watsor_amd.runtime import device_count
        from watsor_amd.detection.hip_gpu import HipObjectDetector
        ndevices = device_count()
    except (ImportError, OSError, RuntimeError):
        return
    if ndevices == 0:
        return

    device = os.environ.get("HIP_DEVICE")
    if device is None:
        try:
            homedir = os.environ.get("HOME")
            assert homedir is not None
            device = open(os.path.join(homedir, ".hip_device")).read().strip()
        except Exception:
            pass

    if device is not None:
        try:
            device = int(device)
        except Exception as e:
            raise TypeError("HIP device number (HIP_DEVICE or ~/.hip_device) must be an integer") from e
        yield device, HipObjectDetector
    else:
        for device in range(ndevices):
            yield device, HipObjectDetector
