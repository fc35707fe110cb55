"""Minimal TTSInterface (the ESPnet contract the reference's drivers rely on; reference call sites
tts.py:356-357, ..._sa.py:605-620): a `reporter` with `.report(list_of_single_key_dicts)`, `add_arguments`,
`forward`, `inference`, `base_plot_keys`.  chainer is not a dependency here: the reporter keeps the last
values (and forwards to an optional observer callable) instead of going through chainer.reporter."""


class Reporter(object):
    def __init__(self):
        self.last = {}
        self.observer = None

    def report(self, dicts):
        for d in dicts:
            self.last.update(d)
        if self.observer is not None:
            self.observer(dict(self.last))


class TTSInterface(object):
    @staticmethod
    def add_arguments(parser):
        return parser

    def __init__(self):
        self.reporter = Reporter()

    def forward(self, *args, **kwargs):
        raise NotImplementedError("forward method is not implemented")

    def inference(self, *args, **kwargs):
        raise NotImplementedError("inference method is not implemented")

    @property
    def base_plot_keys(self):
        return list()
