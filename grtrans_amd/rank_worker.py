"""Rank r > 0 of `grtrans().run_pgrtrans(ngpus=N, ...)`: the same grtrans_main call as rank 0 (module pgrtrans reads its
rank from GRT_RANK / GRT_WORLD_SIZE / GRT_LOCAL_RANK / GRT_COMM_FILE), results discarded -- rank 0 holds the image."""
import pickle
import sys

from .grtrans_batch import grtrans, grtrans_inputs
from .pgrtrans import pgrtrans


def main(path):
    kwargs = pickle.load(open(path, "rb"))
    x = grtrans()
    x.inputs = grtrans_inputs(**kwargs)
    x._call_grtrans_main(pgrtrans)
    pgrtrans.del_pgrtrans_data()


if __name__ == "__main__":
    main(sys.argv[1])
