from .sharding import shard_range  # noqa: F401
