"""Stand-in for espnet.nets.tts_interface (ESPnet v0.8): TTSInterface sets `self.reporter`.

ESPnet's Reporter is a chainer.Chain whose report(dicts) forwards each single-key dict to
chainer.reporter.report. chainer is absent here; this one just keeps the last report list.
"""


class Reporter(object):
    def __init__(self):
        self.last = []

    def report(self, dicts):
        self.last = list(dicts)


class TTSInterface(object):
    @staticmethod
    def add_arguments(parser):
        return parser

    def __init__(self):
        self.reporter = Reporter()

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def inference(self, *args, **kwargs):
        raise NotImplementedError

    @property
    def base_plot_keys(self):
        return list()
