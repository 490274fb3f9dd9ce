import os


def makedirs(path):
    os.makedirs(path, exist_ok=True)
