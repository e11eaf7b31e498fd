"""MI355X-native hot path of the neurst speech-translation models."""
import os


def want_hw_queues(n=6):
    """HIP gives a process 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise, and streams that share a hardware
    queue serialise.  A train step here keeps 3-4 streams busy (main, CTC, weight gradients, communication) and RCCL adds
    its own.  Measured on MI355X (LUT step, one rank under torch.distributed.run; without / with the bucketed all-reduce
    path forced on): 4 queues 17.4 / 17.8 ms, 5: 16.5 / 17.2, 6: 16.6 / 16.9, 8: 16.5 / 22.3 (more queues than the step has
    streams is NOT harmless).  The variable is read when the HIP runtime starts, so this must run before the first
    torch.cuda call of the process."""
    try:
        cur = int(os.environ.get('GPU_MAX_HW_QUEUES', '') or 0)
    except ValueError:
        cur = 0
    if cur < n:
        os.environ['GPU_MAX_HW_QUEUES'] = str(n)


# Importing the package changes nothing in the process: the programs that train (bench.py, st/bin/run_lut.py, the tests'
# conftest) call want_hw_queues() themselves, first thing.  NEURST_HW_QUEUES in the environment asks for it at import.
if os.environ.get('NEURST_HW_QUEUES'):
    want_hw_queues(int(os.environ['NEURST_HW_QUEUES']))
