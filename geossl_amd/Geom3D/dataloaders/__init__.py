from .dataloaders_AtomTuple import (AtomTupleExtractor, BatchAtomTuple, Data,  # noqa: F401
                                    DataLoaderAtomTuple)
