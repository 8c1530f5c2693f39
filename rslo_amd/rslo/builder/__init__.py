
from rslo import extend_path as _extend_path

_extend_path(__path__, __name__)
