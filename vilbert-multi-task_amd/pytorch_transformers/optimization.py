from vilbert.optim import AdamW, ConstantLRSchedule, WarmupConstantSchedule, WarmupLinearSchedule  # noqa: F401
