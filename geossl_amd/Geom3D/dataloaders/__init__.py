from .dataloaders_AtomTuple import AtomTupleExtractor, BatchAtomTuple  # noqa: F401
