"""Iteration observer (drop-in for nsol/observer.py:21-161): keeps a host copy
of every iterate the solver reports and evaluates user measures afterwards.
Attaching an observer forces one device-to-host copy per iteration."""
import numpy as np


class Observer(object):

    def __init__(self, name="Observer"):
        self._name = name
        self._x_list = []
        self._measures = []
        self._measures_names = []
        self._dic_measures = {}
        self._computational_time = None

    def add_x(self, x):
        self._x_list.append(x)

    def set_name(self, name):
        self._name = name

    def get_name(self):
        return self._name

    def clear_x_list(self):
        self._x_list = []

    def get_x_list(self):
        return self._x_list

    def set_measures(self, measures_dic):
        for name, fn in measures_dic.items():
            self._measures_names.append(name)
            self._measures.append(fn)
            self._dic_measures[name] = None

    def get_measures(self):
        return self._dic_measures

    def set_computational_time(self, computational_time):
        self._computational_time = computational_time

    def get_computational_time(self):
        return self._computational_time

    def compute_measures(self):
        for name, fn in zip(self._measures_names, self._measures):
            self._dic_measures[name] = np.array(
                [fn(x) for x in self._x_list], dtype=float)
