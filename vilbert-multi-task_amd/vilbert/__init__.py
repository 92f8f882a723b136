"""MI355X-native ``vilbert`` package: import-compatible with the reference's ``vilbert`` for the model
path (``from vilbert.vilbert import BertConfig, BertForMultiModalPreTraining, VILBertForVLTasks``)."""
