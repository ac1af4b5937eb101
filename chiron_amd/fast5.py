"""Minimal fast5 (HDF5) reader -- placeholder; see DESIGN.md ("next")."""


def read_fast5(path):
    raise NotImplementedError("fast5 reading is not built yet; feed .signal files (chiron_eval accepts them too)")


def read_raw_signal(path):
    return read_fast5(path)[0]["signal"]
