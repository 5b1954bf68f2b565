"""The fields of BaseLearner.__init__ that the retrieval learner uses (methods/base.py:15-28); the exemplar-memory / NME
machinery of PyCIL is outside the hot path (memory_size is 0 in every LPI config)."""
import numpy as np


class BaseLearner(object):
    def __init__(self, args):
        self._cur_task = -1
        self._known_classes = 0
        self._total_classes = 0
        self._network = None
        self._old_network = None
        self._data_memory, self._targets_memory = np.array([]), np.array([])
        self.topk = 5
        self._memory_size = args['memory_size']
        self._memory_per_class = args['memory_per_class']
        self._fixed_memory = args['fixed_memory']
        self._device = args['device'][0]
        self._multiple_gpus = args['device']

    @property
    def exemplar_size(self):
        return len(self._data_memory)
