# -*- coding:utf-8 -*-
"""Process-wide name counters for repeated net blocks (deeptables/utils/counter.py:3-9)."""
_data_ = {}


def next_num(counter_name):
    _data_[counter_name] = _data_.get(counter_name, -1) + 1   # indices begin at 0
    return _data_[counter_name]


def reset(counter_name=None):
    if counter_name is None:
        _data_.clear()
    else:
        _data_.pop(counter_name, None)
