"""Like the reference's modules/file_utils.py:20-21, this module imports boto3 / botocore at import time."""
import os

import boto3                                   # noqa: F401  (not installed on the GPU box: run_univl_amd.py stubs it)
from botocore.exceptions import ClientError    # noqa: F401

PYTORCH_PRETRAINED_BERT_CACHE = os.path.join(os.path.expanduser("~"), ".pytorch_pretrained_bert")
