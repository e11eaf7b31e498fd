"""MI355X-native hot path of the neurst speech-translation models."""
import os


def want_hw_queues(n=6):
    """HIP gives a process 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and streams that share a hardware
    queue serialise.  A train step here keeps 3-4 streams busy (main, CTC, weight gradients, communication) and RCCL adds
    its own.  Measured on MI355X (LUT step, one rank under torch.distributed.run; without / with the bucketed all-reduce
    path forced on): 4 queues 17.4 / 17.8 ms, 5: 16.5 / 17.2, 6: 16.6 / 16.9, 8: 16.5 / 22.3 (more queues than the step has
    streams is NOT harmless).  The variable is read when the HIP runtime starts, so this must run before the first
    torch.cuda call of the process (importing this package first does that); NEURST_HW_QUEUES overrides the 6."""
    try:
        cur = int(os.environ.get('GPU_MAX_HW_QUEUES', '') or 0)
    except ValueError:
        cur = 0
    if cur < n:
        os.environ['GPU_MAX_HW_QUEUES'] = str(n)


want_hw_queues(int(os.environ.get('NEURST_HW_QUEUES', '6')))
