"""The TTSInterface the plug-in classes derive from (the ESPnet contract the reference's drivers rely on; reference call sites
tts_train.py:384 `assert issubclass(model_class, TTSInterface)`, tts.py:360,620 and tts_distill.py:378,642 `assert isinstance(model,
TTSInterface)`, ..._sa.py:605-620 `self.reporter.report([...])`).

When ESPnet is importable, `TTSInterface` here IS a subclass of `espnet.nets.tts_interface.TTSInterface`, so those asserts pass on the
reference's own drivers and `reporter.report(dicts)` is forwarded to ESPnet's reporter (a chainer.Chain that hands each single-key dict to
chainer.reporter.report: LogReport / PlotReport see the named losses).  Without ESPnet (this build's own drivers) the same class stands alone.
Either way the reporter keeps the last values (`.last`) and forwards to an optional observer callable."""

try:  # the reference's environment
    from espnet.nets.tts_interface import TTSInterface as _EspnetTTSInterface
except Exception:  # ESPnet (or its chainer dependency) is not installed: stand-alone interface
    _EspnetTTSInterface = None

ESPNET_BASE = _EspnetTTSInterface  # None, or the ESPnet class the plug-ins are instances of


class Reporter(object):
    def __init__(self, upstream=None):
        self.last = {}
        self.observer = None
        self.upstream = upstream  # ESPnet's Reporter (chainer.Chain) when ESPnet is installed

    def report(self, dicts):
        dicts = list(dicts)
        for d in dicts:
            self.last.update(d)
        if self.upstream is not None:
            self.upstream.report(dicts)
        if self.observer is not None:
            self.observer(dict(self.last))


class TTSInterface(*(() if _EspnetTTSInterface is None else (_EspnetTTSInterface,))):
    @staticmethod
    def add_arguments(parser):
        return parser

    def __init__(self):
        upstream = None
        if _EspnetTTSInterface is not None:
            _EspnetTTSInterface.__init__(self)  # sets self.reporter = ESPnet's Reporter
            upstream = self.__dict__.get("reporter", getattr(self, "reporter", None))
        self.__dict__["reporter"] = Reporter(upstream)  # plain attribute: never registered as a torch sub-module / chainer link of the model

    def forward(self, *args, **kwargs):
        raise NotImplementedError("forward method is not implemented")

    def inference(self, *args, **kwargs):
        raise NotImplementedError("inference method is not implemented")

    @property
    def base_plot_keys(self):
        return list()
