from .cars import CARS

__all__ = ["CARS"]
