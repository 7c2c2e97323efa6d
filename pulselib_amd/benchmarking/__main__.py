from .poker import main

raise SystemExit(main())
