from vilbert.distributed import DistributedDataParallel  # noqa: F401
