"""JSON encoder for Path and numpy scalars (reference ``src/orcAI/json_encoder.py``)."""

import json
from pathlib import Path

import numpy as np


class JsonEncoderExt(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, Path):
            return str(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        return super().default(obj)
