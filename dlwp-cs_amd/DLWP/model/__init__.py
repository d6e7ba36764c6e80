from .models import DLWPFunctional   # noqa: F401
