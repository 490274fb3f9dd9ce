from .dataloaders_AtomTuple import (AtomTupleExtractor, BatchAtomTuple, Data,  # noqa: F401
                                    DataLoaderAtomTuple)
from .device_dataset import DatasetBatch, DeviceDataset, DeviceLoader  # noqa: F401
