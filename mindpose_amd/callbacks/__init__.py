from .eval_callback import EvalCallback  # noqa: F401
